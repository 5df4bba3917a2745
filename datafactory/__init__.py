"""Host-side data plumbing for the drop-in drivers (reference datafactory/)."""
