"""Shared constants for the parity tests (must match tests/golden/make_golden.py)."""
SMALL_VOCAB = dict(concept=60, token=70, predictable_token=50, relation=26, concept_char=20, token_char=22)
SMALL_GEN_ARGS = (8, 12, 8, 12, [(3, 16)], 10, 10, 6, 8, 2)   # char/word dims, filters, rel_dim, rnn
