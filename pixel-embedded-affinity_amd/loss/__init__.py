"""Mirror of the reference's `loss` package for the hot path (same module and function names)."""
