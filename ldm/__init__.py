"""Drop-in name shim: the reference's ``ldm.*`` dotted paths (YAML ``target:`` strings, ``inference.py`` imports)
resolve to the MI355X-native implementation in ``instancediffusion_amd``.  No code lives here."""
