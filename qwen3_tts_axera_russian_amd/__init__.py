"""MI355X-native Qwen3-TTS hot path: HIP kernels behind the reference's C / socket boundaries.

Only what the path needs lives here: csrc/ (gfx950 kernels + the C ABI of include/*.h) and the
host-side mirror of the reference's front-end (bindings, the three socket servers, the client).
"""
import os

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
LIB_PATH = os.path.join(LIB_DIR, "libqwen3tts.so")
TEST_LIB_PATH = os.path.join(LIB_DIR, "libqwen3tts_test.so")   # kernel-level hooks for tests/ and bench.py only
