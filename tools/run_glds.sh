SH="256,256,3,1,20 128,128,3,1,40 64,128,3,2,160 128,256,3,2,80 256,512,3,2,40 128,128,3,2,80 256,256,3,2,40 768,512,1,1,20 1024,512,1,1,20 512,256,1,1,40 768,256,1,1,40 512,512,1,1,20"
echo "== rows + glds 128x128x2"; python tools/bench_conv.py --batch 128 --halo 0 $SH
echo "== rows + glds 256x128x3"; DYOLO_GLDS_BIG=1 python tools/bench_conv.py --batch 128 --halo 0 $SH
