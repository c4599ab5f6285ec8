#!/bin/bash
# Builder's probe: tools/prop_az.py over synthetic_600x1000_64 at a sparse threshold, one image per search and with
# --batch-images 8 / 16 (second run of each: MIOpen's search and the shape's plans done) -- seconds per image as the tool reports.
cd "$(dirname "$0")/../../az-net_amd/tools"
for nb in 1 8 16; do
  for rep in 1 2; do
    python prop_az.py --net synthetic --imdb synthetic_600x1000_64 --tz ${1:-0.4353} --tune-backbone --batch-images $nb --exp cli_nb_$nb --def x --def_fc y > /tmp/cli_$nb.log 2>&1
  done
  echo "batch-images $nb: $(grep 'average proposal generation time' /tmp/cli_$nb.log)"
done
rm -rf ../output/cli_nb_*
