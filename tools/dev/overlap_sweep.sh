for cfg in "-1 1 1" "-1 4 4" "-1 8 4" "-1 32 4" "-1 128 4" "0 8 1" "1 8 1" "0 32 1" "1 32 1" "1 2 2"; do set -- $cfg
  steps=$(( 1024 / $2 )); [ $steps -lt 6 ] && steps=6; [ $steps -gt 64 ] && steps=64
  out=$(timeout 300 python bench.py --spp $2 --streams $3 --overlap $1 --steps $steps --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
  echo "overlap $1 spp $2 streams $3: $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step')")"
done
