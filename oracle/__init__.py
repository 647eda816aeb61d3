"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see gsr_oracle.c header)."""
