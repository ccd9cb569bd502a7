out=gpurun_out/r3c8; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_optim.py tests/test_gpu_dp_training.py -q -m gpu > $out/tests.log 2>&1; tail -4 $out/tests.log
timeout -k 10 300 python bench.py --config c5 --train 2>/dev/null | cut -c1-330
timeout -k 10 300 python bench.py --train --batch 64 2>/dev/null | cut -c1-330
