"""Host-side helpers of the callers (mirror of the reference's ``utils`` package)."""
