set -e
mkdir -p gpurun_out/suite
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/suite/test.log 2>&1 || { tail -40 gpurun_out/suite/test.log; exit 1; }
tail -2 gpurun_out/suite/test.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
