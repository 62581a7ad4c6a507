#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
echo "rc=$?"
tail -6 gpurun_out/smoke.log
