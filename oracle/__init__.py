"""CPU oracle (test infrastructure only) — see oracle/nplda_oracle.py."""
