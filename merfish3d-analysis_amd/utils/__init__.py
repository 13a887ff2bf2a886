"""L3 numeric utilities of the hot path (mirror of the reference's ``utils`` package)."""
