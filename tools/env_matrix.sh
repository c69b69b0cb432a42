cd $GRAFT_REPO_ROOT
for e in "ORBX_SIDE_BLUR=3" "ORBX_SIDE_BLUR=2 ORBX_EARLY_FAST=2" "ORBX_EARLY_FAST=1 ORBX_RESIZE_LDS=2" "ORBX_SIDE_BLUR=0 ORBX_RESIZE_LDS=2 ORBX_EARLY_FAST=2" "ORBX_STREAMS=3" "ORBX_BLUR=valu ORBX_FAST_VARIANT=1"; do
  env $e python -m pytest tests/test_extractor_gpu.py -m gpu -x -q -k "batch or bench or seeded or random" > /tmp/t.log 2>&1; echo "$e: $(tail -1 /tmp/t.log)"
done
