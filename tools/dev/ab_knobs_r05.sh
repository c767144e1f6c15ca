# developer A/B (round 5, end of round): the split-K / tile-selection knobs re-swept on the final C2 step, one box, interleaved
# baselines.  Needs the developer build (SEDT_DEV_BUILD=1 python -m sound_event_detection_transformer_amd._build).
export SEDT_DEV=1 SEDT_LIB_AB=build/dev/libsedt_hip_dev.so
run() {
  echo -n "$1: "
  env $1 python bench.py --no-cpu-baseline --no-kernels --no-other-configs --no-families --steps 150 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
}
run X=0
run SEDT_SPLITK_TARGET=256
run SEDT_SPLITK_TARGET=512
run SEDT_SPLITK_TARGET_WIDE=48
run SEDT_SPLITK_TARGET_WIDE=96
run X=0
run SEDT_WGRAD4_MIN=128
run SEDT_WGRAD4_MIN=512
run SEDT_WGRAD_KSLICE=1
run SEDT_WGRAD4_BIAS=1
run X=0
run SEDT_IGEMM3_W16_TILES=256
run SEDT_IGEMM3_W16_MINK=1024
run SEDT_IGEMM3_STAGES=3
run SEDT_ADAMW_NT=0
run X=0
