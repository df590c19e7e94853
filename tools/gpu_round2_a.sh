cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2_agrees" 2>&1 | tail -5
timeout 300 python tools/op_profile2.py 256 > gpurun_out/op2_b256.txt 2>&1; cat gpurun_out/op2_b256.txt
timeout 300 python tools/op_profile2.py 512 2 > gpurun_out/op2_b512_t2.txt 2>&1; cat gpurun_out/op2_b512_t2.txt | head -5; grep totals gpurun_out/op2_b512_t2.txt
timeout 300 python tools/op_profile2.py 512 1 > gpurun_out/op2_b512_t1.txt 2>&1; cat gpurun_out/op2_b512_t1.txt | head -3; grep totals gpurun_out/op2_b512_t1.txt
