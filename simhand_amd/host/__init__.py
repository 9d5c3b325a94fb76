"""Host-side mirror of the reference's Python interface for the contrastive
pre-training hot path (src/models/*, src/experiments/main.py surface)."""
