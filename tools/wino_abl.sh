cd $GRAFT_REPO_ROOT
for d in 0 2 4 6 16 18 22; do echo "debug $d"; DODT_CONV_DEBUG=$d timeout -k 10 100 python3 tools/conv_bench.py 10 f32; done
