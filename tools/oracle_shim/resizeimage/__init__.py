"""Empty stand-in so `from resizeimage import resizeimage` succeeds (container-only, test tooling)."""
class _Absent:
    def __getattr__(self, name):
        raise AttributeError("python-resize-image is absent from this image (oracle shim stub)")
resizeimage = _Absent()
