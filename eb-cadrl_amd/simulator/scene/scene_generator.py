from ebcsim.scene import SceneConfig, generate_scene, load_scene, save_scene  # noqa: F401  (simulator/scene/scene_generator.py)
