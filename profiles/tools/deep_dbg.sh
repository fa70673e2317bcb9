cd $GRAFT_REPO_ROOT
for D in 0 1 3 4 8 16 19 31; do
  echo "=== DLV_DEEP_DBG=$D"
  DLV_ALLOW_WRONG_RESULTS=1 DLV_DEEP_DBG=$D python3 profiles/zreg_ab.py 0 2 128,256,1024 fp16 2>&1 | grep -E "conv3_deep" | grep -v "^{"
done
