for rb in 1 2 4 8; do
  echo "== EMCID_LINEAR_RB=$rb"
  EMCID_LINEAR_RB=$rb MB_TUNE=0 timeout -k 10 200 python scripts/mb_linear.py 2>&1 | python -c "
import sys
for line in sys.stdin:
    parts=[p.strip() for p in line.split('|')]
    if len(parts)<5: continue
    keep=[p for p in parts[2:-2] if p.startswith('128x128/4w/pf2') or p.startswith('160x128/8w/pf2') or p.startswith('auto')]
    print(parts[0][:26],'|',' | '.join(keep))
"
done
