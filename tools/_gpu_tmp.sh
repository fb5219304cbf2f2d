set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_r06_box.sh E 2>&1 | tail -14
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "box_probe or b32_full_vs_oracle" 2>&1 | tail -4
