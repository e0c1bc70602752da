"""CPU oracle for the Qwen3-TTS hot path -- test infrastructure only (see q3_oracle.c)."""
