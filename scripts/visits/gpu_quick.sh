cd ${GRAFT_REPO_ROOT:-.}; export HSA_ENABLE_IPC_MODE_LEGACY=0; timeout 600 python -m pytest tests/test_gpu_estimator.py -m gpu -q -p no:cacheprovider -k "gather or communicator" 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -30
python bench.py --gpus 1 --check; echo "check rc=$?"
