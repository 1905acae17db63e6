cd /root/repo
python - <<'PY'
import os, sys, subprocess, shutil, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as G, desamba_amd as D
G.demo_dir()
d = "data/bench_strain"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
subprocess.run([sys.executable, "tools/synth_ref.py", d + "/syn.fa", "320", "11", "3", "60", "12"], check=True, stderr=subprocess.DEVNULL)
D.build_index(d + "/syn.fa", d + "/index"); os.remove(d + "/syn.fa")
PY
G=desamba_amd/bin/deSAMBA; I=data/bench_strain/index
python tools/gen_fastq.py $I /dev/shm/st.fq 131072 50000 0.15 1000 ont 16
for r in 0 1 2; do echo "== strain ramp $r"; for rep in 1 2; do DSB_CLI_RAMP=$r DSB_CLI_TRACE=1 $G classify $I /dev/shm/st.fq -o /dev/shm/st.sam 2>&1 | grep -E "processed|^\[gpu" | cut -c1-200; done; done
rm -f /dev/shm/st.fq /dev/shm/st.sam; rm -rf data/bench_strain
