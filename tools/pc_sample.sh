#!/bin/bash
# PC sampling of the headline rollout kernel (rocprofv3 beta feature): where the critical wave spends its cycles.
# usage (through gpurun, repo root): tools/pc_sample.sh [method=host_trap|stochastic]
M=${1:-host_trap}
R=$PWD; O=$R/gpurun_out/pcs_$M; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
if [ "$M" = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; else U="--pc-sampling-unit time --pc-sampling-interval 1"; fi
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M $U -d $O -o pcs --output-format csv -- \
  python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/run.log 2>&1
echo rc=$? >> $O/run.log
find $O -type f | head -20
for f in $(find $O -name "*.csv"); do echo == $f; wc -l $f; head -3 $f | cut -c1-300; done
# keep the merge small: aggregate (code object, offset) counts
python - <<PY
import csv, glob, collections, json, os
out = {}
for f in glob.glob("$O/**/*pc_sampling*.csv", recursive=True):
    cnt = collections.Counter()
    with open(f) as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        for row in rd:
            key = tuple(row.get(c, "") for c in cols if c.lower() in ("code_object_id", "code_object_offset", "instruction", "instruction_comment", "stall_reason", "wave_issued", "instruction_type", "no_issue_reason"))
            cnt[key] += 1
    out[os.path.basename(f)] = {"columns": cols, "rows": sum(cnt.values()), "top": [[list(k), v] for k, v in cnt.most_common(4000)]}
    os.remove(f)
json.dump(out, open("$O/agg.json", "w"))
print({k: v["rows"] for k, v in out.items()})
PY
tail -5 $O/run.log
