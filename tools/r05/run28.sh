#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
SECONDS=0
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
echo "suite seconds $SECONDS"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
