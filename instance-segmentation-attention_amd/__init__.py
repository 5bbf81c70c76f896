"""MI355X-native ReSeg hot path (drop-in for code/lib/archs/reseg.py of the reference)."""
__all__ = ["lib", "engine", "network"]
