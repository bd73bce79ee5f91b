"""CPU checker (test infrastructure only): see oracle/README.md."""
