set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest.txt 2>&1; tail -3 gpurun_out/r06/gputest.txt
bash tools/profile_round.sh r06 > gpurun_out/r06/profile_round.log 2>&1
bash tools/pmc_contract.sh > gpurun_out/r06/pmc_contract_run.log 2>&1
python tools/reference_contract.py --sizes 32,64,128,256,512,1024,2048,4096 > gpurun_out/r06/reference_contract.txt 2>&1
python tools/convolution_example.py > gpurun_out/r06/convolution_example.txt 2>&1
python tools/readme_table.py gpurun_out/r06/readme_table.md > gpurun_out/r06/readme_table.log 2>&1
find gpurun_out -name "*.db" -delete
du -sh gpurun_out
