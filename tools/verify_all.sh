#!/bin/bash
# Everything the round-end driver runs, in one go: build, CPU tests here, then smoke + GPU tests +
# the default bench on a GPU box (through gpurun).  usage: tools/verify_all.sh
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" | tail -1
python -m pytest tests -x -q -m "not gpu" | tail -1
/usr/local/graft/bin/gpurun --timeout 1100 -- 'timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1 && timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -2 && timeout -k 10 200 python bench.py 2>/dev/null | head -c 400' 2>&1 | tail -4
