for rep in 1 2 3; do for ps in 0 1; do
  for idx in cfg5shape_8192streams_barsonly cfg4_n16384_bars; do
    r=$(WF_HIP_LIB=variants/lib_dev.so WF_HIP_BAR_PS=$ps python tools/shape_bench.py $idx 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(r['frac'],4), round(r['frac_events'],4))")
    echo "shape $idx PS=$ps: $r"
  done
done; done
