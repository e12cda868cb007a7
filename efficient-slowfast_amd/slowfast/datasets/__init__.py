"""Input step either side of the hot path (reference slowfast/datasets/utils.py, transform.py)."""
