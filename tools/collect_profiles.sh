#!/bin/bash
# gpurun_out/<tag>/ (what tools/round_end_bench.sh wrote) -> profiles/<tag>_*: top-level files as they are, each PMC directory
# as ONE file (its rocpd_summary tables under "#### <pass>" headers), profiles/traffic.json replaced by the regenerated one.
#   usage: collect_profiles.sh <tag>
TAG=$1; R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/$TAG; P=$R/profiles
for f in $O/*.json $O/*.txt; do
  b=$(basename $f)
  [ $b = traffic.json ] && continue
  [ -s $f ] && cp $f $P/${TAG}_$b
done
for d in $O/pmc_*; do
  out=$P/${TAG}_$(basename $d).txt; : > $out
  for t in $d/*.txt; do echo "#### $(basename $t .txt)" >> $out; cat $t >> $out; done
done
grep -q "Traceback\|matches" $O/traffic.log || cp $O/traffic.json $P/traffic.json
ls $P | grep -c "^${TAG}_"
