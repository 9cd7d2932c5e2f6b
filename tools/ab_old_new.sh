# usage (through gpurun, from the repo root): [CMD="python tools/iter_anatomy.py | head -1"] bash tools/ab_old_new.sh
# Same-box A/B of two whole trees.  Before the call, in the build container:
#   rm -rf _ab_old; mkdir _ab_old; git archive <commit> | tar -x -C _ab_old; make -C _ab_old/vlgp_amd/csrc -j8
# (_ab_old/ is git-ignored and travels with the snapshot).  Default: bench.py's value, the SURVEY-protocol value and the
# phase times, alternating three times; with CMD that command in both trees instead.
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d.get("value_survey_protocol"), d.get("phase_ms_per_step"))'
for i in 1 2 3; do
  if [ -n "${CMD:-}" ]; then
    (cd _ab_old && echo -n "old " && bash -c "$CMD")
    echo -n "new " && bash -c "$CMD"
  else
    (cd _ab_old && python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "$show" old)
    python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "$show" new
  fi
done
