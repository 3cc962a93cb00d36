cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  MVPTR_LIB=diag python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('default order', d['ms_per_step'])"
  MVPTR_LIB=diag MVPTR_NT_GROUP=4,6 python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('4,6          ', d['ms_per_step'])"
  MVPTR_LIB=diag MVPTR_NT_GROUP=4,4 python3 bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('4,4          ', d['ms_per_step'])"
done
MVPTR_LIB=diag python3 bench.py --no-extras --steps 20 --warmup 5 --fixed-length 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fixed default', d['ms_per_step'])"
MVPTR_LIB=diag MVPTR_NT_GROUP=4,6 python3 bench.py --no-extras --steps 20 --warmup 5 --fixed-length 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fixed 4,6    ', d['ms_per_step'])"
