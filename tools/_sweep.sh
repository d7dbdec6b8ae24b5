for st in 6 8 11 16 32; do
  echo "BA_LIN_STEPS=$st"
  BA_LIN_STEPS=$st python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
