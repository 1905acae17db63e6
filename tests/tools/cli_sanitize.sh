#!/bin/bash
# The CLI's reader (mapped waves, parallel pieces, BGZF / gzip inflaters, pipes, the FASTA compat mode) built with
# AddressSanitizer + UBSan and run through its parser tests.     tests/tools/cli_sanitize.sh
cd "$(dirname "$0")/../.."
DSB_HARNESS_CFLAGS="-fsanitize=address,undefined -fno-sanitize-recover=undefined -g" ASAN_OPTIONS=detect_leaks=0 python3 -m pytest tests/test_cli_parser.py -x -q
