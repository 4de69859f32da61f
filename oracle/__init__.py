"""CPU oracle (test infrastructure). See snake_oracle.c for the contract."""
