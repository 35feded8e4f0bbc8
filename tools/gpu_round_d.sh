mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_bf16_storage.py -x -q -k "stem_backward_fused" > gpurun_out/t_stem.log 2>&1; echo "stem rc=$?"; tail -5 gpurun_out/t_stem.log
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q > gpurun_out/t_cfg.log 2>&1; echo "configs rc=$?"; tail -3 gpurun_out/t_cfg.log
bash tools/ab_env.sh gpurun_out/ab_stem_cfg3.txt 3 "--tune-file profiles/r4_cfg3_tune.json --dtype bf16 --batch 128 --image-size 512" "LOANS_STEM_BWD_FUSED=1" "LOANS_STEM_BWD_FUSED=0"
bash tools/ab_env.sh gpurun_out/ab_stem_r50.txt 2 "--tune-file profiles/r4_r50_tune.json --dtype bf16 --batch 64 --image-size 512 --resnet50" "LOANS_STEM_BWD_FUSED=1" "LOANS_STEM_BWD_FUSED=0"
