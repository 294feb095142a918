SH="256,256,3,1,20 128,128,3,1,40 64,128,3,2,160 128,256,3,2,80 256,512,3,2,40 64,64,3,2,160 128,128,3,2,80 256,256,3,2,40 768,512,1,1,20 1024,512,1,1,20 512,256,1,1,40 768,256,1,1,40 512,512,1,1,20"
B=${1:-256}
echo "== persistent"; python tools/bench_conv.py --batch $B --halo 0 $SH
echo "== non-persistent"; DYOLO_GLDS_BIG=2 python tools/bench_conv.py --batch $B --halo 0 $SH
