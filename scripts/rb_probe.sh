for rb in 1 2 4 8 16 50; do echo "rb $rb"; EMCID_SP16_RB=$rb python scripts/mb_linear_sp16.py 2>&1 | head -5 | cut -d'|' -f1,5 ; done
