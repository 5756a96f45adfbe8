"""CPU oracle of the sclens() hot path: TEST INFRASTRUCTURE ONLY (see sclens_oracle.py)."""
