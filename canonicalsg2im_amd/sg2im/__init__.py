"""MI355X-native stand-ins for the reference's `sg2im` package (hot-path modules only)."""
