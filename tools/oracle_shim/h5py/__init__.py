"""Empty stand-in so `import h5py` succeeds (container-only, test tooling)."""
def __getattr__(name):
    raise AttributeError(f"h5py.{name}: h5py is absent from this image (oracle shim stub)")
