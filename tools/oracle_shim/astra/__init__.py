"""Empty stand-in so `import astra` in the reference succeeds (container-only, test tooling).
ASTRA is not installable here; nothing in it is emulated — any call raises."""
def __getattr__(name):
    raise AttributeError(f"astra.{name}: astra-toolbox is absent from this image (oracle shim stub)")
