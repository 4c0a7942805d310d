"""Drop-in name shim for the two hot-path functions of the reference's ``utils/model.py``."""
