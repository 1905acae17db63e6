#!/bin/bash
# The CLI's reader (mapped waves, parallel pieces, BGZF / gzip inflaters, pipes, the FASTA compat mode) built with
# AddressSanitizer + UBSan and run through its parser tests.     tests/tools/cli_sanitize.sh
cd "$(dirname "$0")/../.."
DSB_HARNESS_CFLAGS="-fsanitize=address,undefined -fno-sanitize-recover=undefined -g" ASAN_OPTIONS=detect_leaks=0 python3 -m pytest tests/test_cli_parser.py -x -q
# ... and with ThreadSanitizer (reader, inflater and piece-parser threads; a report makes the harness exit 66 and the test fail)
DSB_HARNESS_CFLAGS="-fsanitize=thread -g" TSAN_OPTIONS="report_signal_unsafe=0" python3 -m pytest tests/test_cli_parser.py -x -q
