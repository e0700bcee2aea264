"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see fusion_ref.py header)."""
