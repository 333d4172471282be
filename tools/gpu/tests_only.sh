mkdir -p gpurun_out/tests
python -m pytest tests -m gpu -x -q > gpurun_out/tests/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/tests/pytest.log
