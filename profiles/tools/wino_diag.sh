for dbg in 0 2 4 6; do
  echo "== DLV_ZREG_DBG=$dbg"; DLV_ZREG_DBG=$dbg DLV_LIB=libdelivr_hip_abl_wino.so DLV_ALLOW_WRONG_RESULTS=1 timeout 300 python profiles/wino_ab.py 2 2>&1 | grep -E "zwino_f16_c32x32_d128|zreg_f16_c32x32_d128|^direct|^winograd"
done
