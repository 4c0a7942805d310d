"""Drop-in name shim for the reference's ``grounding_input`` package."""
