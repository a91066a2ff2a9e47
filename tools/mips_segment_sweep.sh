export PYTHONPATH=$PWD
for seg in 8192 15872; do for gr in 8 16 32; do
  echo "SEG0=$seg GROWTH=$gr"
  EMDR2_MIPS_SEG0=$seg EMDR2_MIPS_GROWTH=$gr python tools/scan_launches.py --exp 21015324 512 | tail -1
  EMDR2_MIPS_SEG0=$seg EMDR2_MIPS_GROWTH=$gr python tools/scan_launches.py --exp 2626916 512 | tail -1
done; done
