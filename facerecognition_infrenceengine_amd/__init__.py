"""MI355X-native detect -> align -> embed -> match engine (HIP C-ABI + Python host)."""
__version__ = "0.1.0"
