for cfg in "768 0" "512 0" "1024 0" "768 1" "768 2" "768 3"; do
set -- $cfg
echo "grid $1 dbg $2: $(SEDT_STEM_GRID=$1 SEDT_STEM_DBG=$2 python tools/dev/time_stem.py 2>/dev/null | grep 'B=64')" >> gpurun_out/t15.log
done
