"""Frame-pair pipeline (filled in below)."""
