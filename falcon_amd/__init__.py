"""falcon's vectorise -> ANN -> DBSCAN spectrum-clustering hot path on MI355X (gfx950)."""
__version__ = "0.1.0+mi355x"
