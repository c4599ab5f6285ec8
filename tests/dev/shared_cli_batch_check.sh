#!/bin/bash
# Builder's check: tools/test_shared.py with and without --batch-images writes the same detections.pkl (one process each).
cd "$(dirname "$0")/../../az-net_amd/tools"
export AZ_BACKBONE_DETERMINISTIC=1
for nb in 1 3; do
  python test_shared.py --net_az synthetic --net_frcnn synthetic --imdb synthetic_600x1000_5 --tz 0.4 --batch-images $nb --exp shared_nb_$nb > /tmp/shared_$nb.log 2>&1 || { tail -5 /tmp/shared_$nb.log; exit 1; }
  grep -c "im_detect:" /tmp/shared_$nb.log
done
python - <<'PY'
import glob, pickle, numpy as np
f = [glob.glob("../output/shared_nb_%d/*/*/detections.pkl" % nb)[0] for nb in (1, 3)]
a, b = [pickle.load(open(x, "rb")) for x in f]
same = all(np.array_equal(x, y) for ca, cb in zip(a, b) for x, y in zip(ca, cb))
print("detections.pkl identical:", same)
PY
rm -rf ../output/shared_nb_*
