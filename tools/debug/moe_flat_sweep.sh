for w in 0 64 128 256; do for s in 0 2 4 8; do
  export EXL2_FLAT_WGS=$w EXL2_FLAT_SPLIT=$s; [ $w = 0 ] && unset EXL2_FLAT_WGS; [ $s = 0 ] && unset EXL2_FLAT_SPLIT
  echo "wgs=$w split=$s $(timeout 120 python tools/moe_bench.py 2>/dev/null | head -1 | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["us_per_layer"])' 2>/dev/null)"
done; done
