"""MI355X-native detect -> align -> embed -> match engine (HIP C-ABI + Python host)."""
__version__ = "0.1.0"

from .face_analysis import Face, FaceAnalysis, FaceEngine  # noqa: E402,F401
from .gallery import GalleryMatcher  # noqa: E402,F401
