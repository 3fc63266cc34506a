for d in 0 1 2 4 7; do echo "dbg=$d"; UENC_WATTN_DBG=$d python tools/wattn_bench.py 2>&1 | grep -E "s1 shift 6|s3 shift 6"; done
