"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/cpu_ref.py header)."""
