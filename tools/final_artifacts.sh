#!/bin/bash
# GPU box: the bench lines the round's documents quote, written under gpurun_out/final_TAG/ (copy into profiles/ as rNN_*.json).
TAG=${1:-r05}
O=gpurun_out/final_$TAG; mkdir -p $O
run() { out=$1; shift; python bench.py "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python -c "
import json,sys
d=json.load(open('$O/$out.json')); r=d.get('roofline') or {}; p=d.get('parity') or {}
print('$out', d.get('value'), d.get('unit'), d.get('ms_per_step'), 'frac', r.get('frac'), 'match', p.get('match_rate'), 'iou_min', p.get('iou_min'))"; }
run bench_default
run bench_train --mode train
run bench_f16x2 --dtype f16x2 --no-train --no-extra --no-cpu-baseline
run bench_x1536_fp16 --model yolov8x-p2-repvgg.yaml --imgsz 1536 --batch 8 --dtype fp16 --no-train --no-sweep
run bench_x1536_fp8 --model yolov8x-p2-repvgg.yaml --imgsz 1536 --batch 8 --dtype fp8 --no-train --no-sweep
run bench_x1536_fp8mixed --model yolov8x-p2-repvgg.yaml --imgsz 1536 --batch 8 --dtype fp8-mixed --no-train --no-sweep
run bench_l1280_fp16 --model yolov8l-p2-repvgg.yaml --imgsz 1280 --batch 8 --dtype fp16 --no-train --no-sweep
