"""TEST INFRASTRUCTURE ONLY: parity oracles (numpy restatement + the reference built unmodified into oracle/_ref)."""
