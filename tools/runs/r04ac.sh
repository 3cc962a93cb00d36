O=gpurun_out/r04ac; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 900 python bench.py > $O/bench.txt 2>&1; tail -1 $O/bench.txt | cut -c1-200
timeout 900 python -m pytest tests -x -q -m gpu -k "attention" > $O/t.txt 2>&1; tail -1 $O/t.txt
