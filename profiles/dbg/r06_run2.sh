# round 6: bit identity, launch times, stamps, then the denoiser / full-size parity tests
cd $GRAFT_REPO_ROOT
TAG=${1:-r06c}
timeout 600 python profiles/dbg/seq2_check.py 8 2>&1 | grep -v "compared\|amdgpu.ids" > gpurun_out/${TAG}_seq2_check.txt
cat gpurun_out/${TAG}_seq2_check.txt
RG_DIAG=1 timeout 300 python profiles/dbg/seq2_stamps.py 64 0 > gpurun_out/${TAG}_seq2_stamps.txt 2>&1
head -4 gpurun_out/${TAG}_seq2_stamps.txt | cut -c1-420
timeout 1500 python -m pytest tests/test_denoiser_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/${TAG}_tests.txt
cat gpurun_out/${TAG}_tests.txt
cp -f gpurun_out/parity_gpu.json gpurun_out/${TAG}_parity_gpu.json
