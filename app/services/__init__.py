"""``app/services`` is an empty package in the reference; this is the slot the
MI355X engine fills.  Everything lives in ``facerecognition_infrenceengine_amd``."""
from facerecognition_infrenceengine_amd.face_analysis import Face, FaceAnalysis, FaceEngine  # noqa: F401
from facerecognition_infrenceengine_amd.gallery import GalleryMatcher  # noqa: F401
from facerecognition_infrenceengine_amd.processor import (CameraProcessor, EmbeddingManager,  # noqa: F401
                                                          FaceRecognitionProcessor, InMemoryStore)
from facerecognition_infrenceengine_amd.camera import CameraManager  # noqa: F401
from facerecognition_infrenceengine_amd.server import create_app  # noqa: F401
