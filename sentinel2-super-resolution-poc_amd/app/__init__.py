"""Drop-in replacements for the reference's `app.cnn_super_resolution`, `app.wow_sr` and
`app.farm_sr` modules (same import paths as server/app/main.py:255,335,1072,1080 uses)."""
