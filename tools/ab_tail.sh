# usage: bash tools/ab_tail.sh  -> eager vs replayed tail, sections of the step (no profiler attached)
R=${GRAFT_REPO_ROOT:-.}
DLIO_TAIL_GRAPH=0 python $R/tools/step_sections.py 2>&1 | tail -5
DLIO_TAIL_GRAPH=1 python $R/tools/step_sections.py 2>&1 | tail -5
