cd /root/repo
python -m pytest tests/test_gpu_ops.py -q -x -k "two_piece" 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py -q -x -k "headline_encoder_gradients_with_the_decisions_pinned or headline_shape_train_forward or stream_overlap" 2>&1 | tail -4
python -m pytest tests/test_gpu_modules.py -q -x -k "fire or Fire" 2>&1 | tail -4
bash tools/rep_ab.sh DLIO_DGRAD1_H2=0 2>&1 | tee gpurun_out/ab_dgrad1_h2.txt
