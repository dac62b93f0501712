for cfg in "4 4096" "4 2048" "4 1024" "4 600" "2 2048" "2 1024" "2 4096"; do set -- $cfg; 
 a=$(BEAT_RR_RY=$1 BEAT_RR_BLOCKS=$2 python bench.py --size 512 --size-z 64 --steps 40 --warmup 10 --cpu-sample 0 --no-front 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(round(d['config']['pde_ms'],3))")
 b=$(BEAT_RR_RY=$1 BEAT_RR_BLOCKS=$2 python bench.py --size 256 --iso --steps 60 --warmup 10 --cpu-sample 0 --no-front 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(round(d['config']['pde_ms'],3))")
 echo "ry=$1 blocks=$2  slab512x512x64 pde_ms=$a   256^3 pde_ms=$b"
done
a=$(BEAT_RR=0 python bench.py --size 512 --size-z 64 --steps 40 --warmup 10 --cpu-sample 0 --no-front 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(round(d['config']['pde_ms'],3))")
b=$(BEAT_RR=0 python bench.py --size 256 --iso --steps 60 --warmup 10 --cpu-sample 0 --no-front 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(round(d['config']['pde_ms'],3))")
echo "classic kernels  slab pde_ms=$a   256^3 pde_ms=$b"
