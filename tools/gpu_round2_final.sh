# Round-2 final evidence (GPU box): evidence run, per-op profiles of the three workgroup shapes, batch sweep, smoke.
cd $GRAFT_REPO_ROOT
bash tools/gpu_round2_evidence.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
timeout 300 python tools/op_profile2.py 256 1 8 > gpurun_out/ev/r02_op_profile_wg0.txt 2>&1; tail -1 gpurun_out/ev/r02_op_profile_wg0.txt
timeout 300 python tools/op_profile2.py 512 2 8 > gpurun_out/ev/r02_op_profile_wg0_b512_t2.txt 2>&1; tail -1 gpurun_out/ev/r02_op_profile_wg0_b512_t2.txt
CDX_UNET2_T=3 timeout 300 python tools/op_profile2.py 768 3 8 > gpurun_out/ev/r02_op_profile_wg0_b768_t3.txt 2>&1; tail -1 gpurun_out/ev/r02_op_profile_wg0_b768_t3.txt
timeout 600 bash tools/gpu_batch_sweep.sh > gpurun_out/ev/r02_batch_sweep.txt 2>&1; tail -12 gpurun_out/ev/r02_batch_sweep.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
