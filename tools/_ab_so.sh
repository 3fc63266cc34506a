# same-box A/B of two builds of the library: $1 = script (+ args) to run under each; the previous build is uenc/libuenc_hip_prev.so
set -e
cd $GRAFT_REPO_ROOT
U=uni-encoder-code_amd/uenc
cp $U/libuenc_hip.so /tmp/new.so
cp $U/libuenc_hip_prev.so $U/libuenc_hip.so
echo "== previous build"; "$@"
cp /tmp/new.so $U/libuenc_hip.so
echo "== new build"; "$@"
echo "== previous build again"; cp $U/libuenc_hip_prev.so $U/libuenc_hip.so; "$@"
cp /tmp/new.so $U/libuenc_hip.so
