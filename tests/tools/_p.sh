cd /root/repo
python -c "import __graft_entry__ as g; g.demo_dir()" > gpurun_out/demo.log 2>&1
G=desamba_amd/bin/deSAMBA; I=data/demo/index
python tools/gen_fastq.py $I /dev/shm/h.fq 65536 12000 0.13 3 pacbio 16
DSB_UPLOAD_TRACE=1 DSB_CLI_TRACE=1 $G classify $I /dev/shm/h.fq -o /dev/shm/h.sam 2>&1 | grep -E "processed|upload\]|^\[gpu|trace\] input" | cut -c1-230
rm -f /dev/shm/h.fq /dev/shm/h.sam
