"""bench.py's secondary legs, one module each; bench.py itself keeps the headline path."""
