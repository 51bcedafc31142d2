O=gpurun_out
JFGPU_MATRIX=xs timeout 2400 python -m pytest -m gpu -q -p no:cacheprovider tests > $O/r06_suite_xs.log 2>&1
timeout 2400 python -m pytest -m gpu -q -x -p no:cacheprovider tests > $O/r06_suite_ref.log 2>&1
grep -E "passed|failed" $O/r06_suite_xs.log $O/r06_suite_ref.log | tail -5
