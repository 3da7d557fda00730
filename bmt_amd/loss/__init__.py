"""MI355X-native drop-in for the reference's ``loss`` package."""
