timeout 600 python -m pytest tests/test_training.py -q -x -k "conv_gradients or split_weight" 2>&1 | tail -5
python devtools/wgrad_time.py
