# Round 6, call 2: is test_finetune_models_unpadded_two_streams flaky?  full suite; GELU-backward stash formats; stash A/B in the step
O=gpurun_out/r06b; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do python3 -m pytest tests/test_model_gpu.py -q -m gpu -k "finetune_models_unpadded" 2>&1 | grep -E "passed|failed|worst gradient" | tail -3; done
python3 -m pytest tests -q -m gpu > $O/pytest.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | tail -15
python3 tools/epi_ablate.py --ms 37748 > $O/epi_ablate.log 2>&1; grep -v amdgpu $O/epi_ablate.log | grep -E "gemm |GELU|region|img-emb|torch" | cut -c1-300
for i in 1 2; do
  python3 bench.py --no-extras --steps 20 --warmup 5 --gelu-stash u8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('u8  ', d['ms_per_step'])"
  python3 bench.py --no-extras --steps 20 --warmup 5 --gelu-stash bf16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bf16', d['ms_per_step'])"
done
python3 bench.py --no-extras --steps 20 --warmup 5 --fixed-length --gelu-stash u8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fixed u8  ', d['ms_per_step'])"
python3 bench.py --no-extras --steps 20 --warmup 5 --fixed-length --gelu-stash bf16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fixed bf16', d['ms_per_step'])"
