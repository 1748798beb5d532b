"""CPU oracle of the rtk hot path -- TEST INFRASTRUCTURE ONLY (see oracle/rtk_oracle.h)."""
