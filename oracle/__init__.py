"""CPU oracle — test infrastructure only (see emcid_oracle.py); never imported by emcid_amd/."""
