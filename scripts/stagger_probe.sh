for st in 0 1 2 3 4 6; do echo "stagger $st"; EMCID_SP16_STAGGER=$st MB_DBG=1 python scripts/mb_linear_sp16.py 2>&1 | head -5 | cut -d'|' -f1,5,7,9,15,16,17 ; done
