"""CPU tests (-m "not gpu": oracle vs golden fixtures, host logic, ABI exports, gloo exchange step) and GPU parity tests (-m gpu)."""
