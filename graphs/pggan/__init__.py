"""reference graphs/pggan/ — BASELINE config 1 (z-space walk on the in-repo PGGAN-256 generator)."""
