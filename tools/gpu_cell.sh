#!/bin/bash
# packed two-cell LSTM update in the row-tile kernel (lstm_common.h lstm_cell_pair): parity + headline
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "lstm2_fc or golden or b32 or bf16 or valu or extra_row or harsh" 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=32 ms/step %.3f alt %.3f lstm %.3f frac %.4f fullband %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['roofline']['fullband_ms'], r['value']))"
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --precision bf16_ih 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('bf16_ih ms/step %.3f alt %.3f lstm %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline']['avg_launch_ms'], r['value']))"
python bench.py --steps 3 --warmup 2 --no-cpu-baseline --seconds 10 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('10s ms/step %.3f alt %.3f lstm %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline']['avg_launch_ms'], r['value']))"
