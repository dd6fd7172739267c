"""Test-infrastructure package: CPU oracle of the reference ops (never imported by the product path)."""
