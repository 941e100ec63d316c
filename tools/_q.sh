cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "mix11 $(python $R/tests/corridor_bench.py 65536 mix11 2>/dev/null | grep -o '"seconds": [0-9.]*') dyn20x(16384) $(python $R/tests/corridor_bench.py 16384 dyn20x 2>/dev/null | grep -o '"seconds": [0-9.]*')"
cd $R && python -m pytest tests/test_corridor.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
