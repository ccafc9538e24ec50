# FFT channel chunk at DPOT-L, batch 16 (DPOT_DFT_CC: 0 = the rule, 16 / 32 forced), two repetitions on one box
for rep in 1 2; do for cc in 0 32; do
  echo "== DPOT-L batch 16, DPOT_DFT_CC=$cc (rep $rep)"
  DPOT_DFT_CC=$cc timeout 600 python bench.py --config L --brief --no-alt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['config']['final_loss'])"
done; done
