"""Import-path compatibility with the reference tree (`models.raymarchers.mvpraymarcher`)."""
