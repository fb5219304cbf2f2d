export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsubnet.py -m gpu -q -x -p no:cacheprovider -k "generic or h320 or fbn6 or h190 or fbh" 2>&1 | tail -4
for h in 320; do for n in 257 2056 8224; do HIDDEN=$h timeout 200 python tools/time_lstm.py $n 128 3 2>&1 | tail -1; done; done | tee gpurun_out/generic_times.txt
