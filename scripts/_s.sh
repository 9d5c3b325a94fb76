timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/final_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
SIMHAND_COMMIT=632d0f6 timeout 1200 bash scripts/refresh_profiles.sh r03 > gpurun_out/refresh.log 2>&1; echo "refresh rc $?"; tail -5 gpurun_out/refresh.log
